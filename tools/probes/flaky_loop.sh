# how often does the in-flight test fail, with one new feature switched off at a time?
cd $GRAFT_REPO_ROOT
for cfg in "" "CFP_HEAD_FUSED=0" "CFP_NO_PE_FUSE=1" "CFP_NO_DIRECT_OUT=1" "CFP_OLD_SUMS=1"; do
  fails=0
  for i in 1 2 3 4 5 6; do
    env $cfg python -m pytest tests/test_forward_gpu.py -q -m gpu -k "batches_in_flight" > /tmp/fl.txt 2>&1 || fails=$((fails+1))
  done
  echo "config [$cfg]: $fails failing runs of 6"
done
