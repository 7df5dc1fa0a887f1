mkdir -p gpurun_out/r06q
for c in 0 16 8 4 2; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --alt-steps 0 --no-configs --no-shards --no-extras --bigvgan-chunk $c > gpurun_out/r06q/chunk_$c.json 2> gpurun_out/r06q/chunk_$c.err
  python -c "
import json; j=json.loads(open('gpurun_out/r06q/chunk_$c.json').read().strip().splitlines()[-1]); print('bigvgan-chunk $c: value', round(j['value'],2), 'ms/step', round(j['ms_per_step'],1), 'bigvgan ms', round(j['stages']['bigvgan_ms_per_step'],1))"
done
