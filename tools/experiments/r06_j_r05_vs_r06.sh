# same-box check that round 6's edits of the env step kernels (continuous action slices in the step loops, assemble_obs's Part parameter) cost the
# table-action forms nothing: round 5's tree (git archive 1adcf77 -> .ab/r05tree, built there) against HEAD, tools/kbench.py
o=$PWD/gpurun_out/r06_j; mkdir -p $o
{
for rep in 1 2; do
 for tree in .ab/r05tree .; do
  echo "== tree $tree (rep $rep)"
  (cd $tree && timeout 600 python tools/kbench.py --worlds 8192 1048576 --agents 4 --spl 1 16 20 2>&1 | grep us_per)
  (cd $tree && timeout 600 python tools/kbench.py --worlds 8192 262144 --agents 10 --spl 1 16 2>&1 | grep us_per)
 done
done
} | tee $o/r05_vs_r06_kbench.txt
