mkdir -p gpurun_out/r03f
for v in base wf_nostore wf_nobuild wf_nopay; do
  if [ $v = base ]; then unset CHD_SPATIAL_LIB; else export CHD_SPATIAL_LIB=$GRAFT_REPO_ROOT/channeld_amd/variants/libchd_$v.so; fi
  timeout -s KILL 120 python bench.py --steps 5 --warmup 3 --no-cpu --latency-steps 0 --serial-ticks > gpurun_out/r03f/bench_$v.json 2> gpurun_out/r03f/bench_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/r03f/bench_$v.json'))['e2e']['wire']
print('$v', d['ms_per_build_all'], d['bytes_per_tick'])"
done
