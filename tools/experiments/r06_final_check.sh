# the round's last check on the clean-built tree: __graft_entry__.smoke(), the whole -m gpu suite (a second box: flakiness), the driver's two command lines
o=$PWD/gpurun_out/r06_final_check; mkdir -p $o
flt() { grep -av "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; }
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | flt | tail -3 ) > $o/smoke.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | flt | tail -6 ) > $o/pytest_gpu.txt
( timeout 600 python bench.py --steps 20 --warmup 5 > $o/bench_k20.json 2> $o/bench_k20.err; echo "rc=$?" ) > $o/bench_rc.txt
( /usr/bin/time -v timeout 900 python bench.py > $o/bench.json 2> $o/bench.err; echo "rc=$?" ) >> $o/bench_rc.txt 2>&1
grep -a "Elapsed" $o/bench.err >> $o/bench_rc.txt
cat $o/smoke.txt $o/pytest_gpu.txt $o/bench_rc.txt
python - <<PY
import json
for f in ("$o/bench_k20.json", "$o/bench.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f.split("/")[-1], "value %.4e ms/step %.5f kernel_us %.2f frac %.4f traffic/moved %s cpu_baseline %s" % (d["value"], d["ms_per_step"], r["kernel_us"], r["frac"], r.get("traffic_over_moved"), d.get("cpu_baseline", {}).get("value")))
PY
