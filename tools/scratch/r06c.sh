set -o pipefail
cd $GRAFT_REPO_ROOT
for v in "RBNN_CONV1_X3=0" "RBNN_CONV1_X3_PW=0" "RBNN_CONV1_X3_PW=1" "RBNN_CONV1_X3_PW=2" "RBNN_CONV1_X3_PW=4" "RBNN_CONV1_X3_PW=8"; do
  echo "=== $v"
  env $v bash tools/prof_quick.sh r06c_$v c5 --points 512 --iters 3 --steps 1 --warmup 1 --no-other-mode 2>&1 | grep -E "conv1_pool|conv2_pool_x3|conv_bwd_dense|^[0-9.e+]+ " | cut -c1-150
done
echo "=== conv (1x28x28)"
for v in "RBNN_CONV1_X3=0" "RBNN_CONV1_X3_PW=0"; do
  env $v bash tools/prof_quick.sh r06c_conv_$v conv --steps 3 --warmup 1 --no-other-mode 2>&1 | grep -E "conv1_pool|conv2_pool_x3|^[0-9.e+]+ " | cut -c1-150
done
