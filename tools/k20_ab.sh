for rep in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'])"
HSA_ENABLE_INTERRUPT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nointr ', d['value'], d['ms_per_step'])"
done
