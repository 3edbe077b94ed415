set -e
mkdir -p gpurun_out/r05u
(timeout 900 python -m pytest tests -m gpu -q -x -k "conv" 2>&1 | grep -vE "^(Saving|Loading|Producing|Evaluating|test accuracy|avg softmax|vanishing|increasing|null|image_idx| === |min = |$)" | tail -15) > gpurun_out/r05u/pytest_conv.log; tail -3 gpurun_out/r05u/pytest_conv.log
bash tools/run_conv_variants.sh "-DRBNN_DENSE_SOFTBAR=0" "-DRBNN_DENSE_SOFTBAR=1" "-DRBNN_DENSE_SOFTBAR=0" "-DRBNN_DENSE_SOFTBAR=1" "-DRBNN_DENSE_SOFTBAR=1 -DRBNN_DENSE_PRIO=0"
cp gpurun_out/abl/conv_variants.log gpurun_out/r05u/softbar_variants.log
rm -f gpurun_out/abl/dense_stamps.log
LEVELS=2 EXTRA="-DRBNN_DENSE_SOFTBAR=0 -DRBNN_DENSE_STAMP_WA=0 -DRBNN_DENSE_STAMP_WB=2" STAMP_WA=0 STAMP_WB=2 bash tools/dense_stamps.sh
LEVELS=2 EXTRA="-DRBNN_DENSE_SOFTBAR=1 -DRBNN_DENSE_STAMP_WA=0 -DRBNN_DENSE_STAMP_WB=2" STAMP_WA=0 STAMP_WB=2 bash tools/dense_stamps.sh
LEVELS=2 EXTRA="-DRBNN_DENSE_SOFTBAR=1 -DRBNN_DENSE_STAMP_WA=4 -DRBNN_DENSE_STAMP_WB=6" STAMP_WA=4 STAMP_WB=6 bash tools/dense_stamps.sh
cp gpurun_out/abl/dense_stamps.log gpurun_out/r05u/softbar_stamps.log
