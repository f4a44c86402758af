for f in "" "--no-graph" "--item-dist grouped" "--item-dist zipf"; do echo "bench $f"; python bench.py --no-cpu-baseline $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'])"; done
python scratch/trainer_bench.py device 2>&1 | tail -3
