cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  python -m pytest tests/test_forward_gpu.py -q -m gpu -k "batches_in_flight" > /tmp/fl.txt 2>&1 || { grep -h "AssertionError: \|^E           assert \|^FAILED" /tmp/fl.txt | cut -c1-120; }
done
echo done
