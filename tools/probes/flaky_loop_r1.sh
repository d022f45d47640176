cd $GRAFT_REPO_ROOT/_r1
fails=0
for i in 1 2 3 4 5 6 7 8; do
  python -m pytest tests/test_forward_gpu.py -q -m gpu -k "batches_in_flight" > /tmp/fl.txt 2>&1 || fails=$((fails+1))
done
echo "round-1 tree: $fails failing runs of 8"
grep -h "AssertionError: \|assert 0" /tmp/fl.txt | head -3
