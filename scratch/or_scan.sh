cd recbole-fairrec_amd/csrc
for r in 32 16 8 4 2; do
  sed -i "s/static constexpr int OUTER_ROWS = [0-9]*;/static constexpr int OUTER_ROWS = $r;/" pfcn.hip
  make 2>&1 | grep -E "error" | head -2
  echo "OUTER_ROWS=$r"; python ../../scratch/bpr_outer_bench.py 2>&1 | tail -1
done
