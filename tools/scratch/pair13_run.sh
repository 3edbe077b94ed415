set -e
mkdir -p gpurun_out/r05w
(timeout 900 python -m pytest tests -m gpu -q -x -k "conv" 2>&1 | grep -vE "^(Saving|Loading|Producing|Evaluating|test accuracy|avg softmax|vanishing|increasing|null|image_idx| === |min = |$)" | tail -15) > gpurun_out/r05w/pytest_conv.log; tail -3 gpurun_out/r05w/pytest_conv.log
bash tools/run_conv_variants.sh "-DRBNN_X3FWD_PAIR13=0" "-DRBNN_X3FWD_PAIR13=1" "-DRBNN_X3FWD_PAIR13=0" "-DRBNN_X3FWD_PAIR13=1"
cp gpurun_out/abl/conv_variants.log gpurun_out/r05w/pair13_variants.log
