set -e
mkdir -p gpurun_out/r05v
(timeout 900 python -m pytest tests -m gpu -q -x -k "conv" 2>&1 | grep -vE "^(Saving|Loading|Producing|Evaluating|test accuracy|avg softmax|vanishing|increasing|null|image_idx| === |min = |$)" | tail -15) > gpurun_out/r05v/pytest_conv.log; tail -3 gpurun_out/r05v/pytest_conv.log
bash tools/run_conv_variants.sh "-DRBNN_DENSE_HALFTAP=0" "-DRBNN_DENSE_HALFTAP=1" "-DRBNN_DENSE_HALFTAP=0" "-DRBNN_DENSE_HALFTAP=1" "-DRBNN_DENSE_HALFTAP=1 -DRBNN_DENSE_PRIO=2" "-DRBNN_DENSE_HALFTAP=1 -DRBNN_DENSE_PRIO=0"
cp gpurun_out/abl/conv_variants.log gpurun_out/r05v/halftap_variants.log
