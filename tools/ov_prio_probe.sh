# A/B of the auxiliary stream's priority in batch pipelining (bench.py --overlap); BADGER_AMD_AUX_PRIO is read by bdg_set_overlap
for rep in 1 2 3; do
for p in 0 1 serial; do
if [ $p = serial ]; then python bench.py --no-graph --no-cpu-baseline --steps 40 > gpurun_out/ov_p.json 2> gpurun_out/ov_p.err
else BADGER_AMD_AUX_PRIO=$p python bench.py --no-graph --overlap --no-cpu-baseline --steps 40 > gpurun_out/ov_p.json 2> gpurun_out/ov_p.err; fi
python3 -c "
import json
x=json.loads(open('gpurun_out/ov_p.json').read().strip().splitlines()[-1])
print('rep $rep prio $p', round(x['ms_per_step'],4), round(x['roofline']['kernel_ms'],4))"
done
done
