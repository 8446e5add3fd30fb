O=gpurun_out/r3b_final; mkdir -p $O
set -u
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; rc=$?; tail -4 $O/tests_all.log
if [ $rc -ge 124 ]; then echo "tests were killed (rc $rc): no further GPU step in this call"; exit $rc; fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -2 $O/bench.err
if [ $rc -ge 124 ]; then echo "bench was killed (rc $rc)"; exit $rc; fi
timeout -k 10 200 python bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_reh.json 2> $O/bench_reh.err; tail -c 600 $O/bench_reh.json
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['traffic'],d['stage_ms']);print({k:(v.get('us_per_step') or v.get('ms_per_step') or v.get('ms') or v.get('s_per_step') or v.get('ms_per_utterance')) for k,v in d['extra'].items() if isinstance(v,dict)})"
