#!/bin/bash
# Scaling curve of bench.py on ONE node: N = 1, 2, 4, 8 ranks (one per GPU, RCCL), weak and strong.
#   weak:   every rank owns a configuration shard (16 trajectories of 64x2x256x256); N = 8 is BASELINE configs[3] itself
#   strong: the configuration's global batch (128 trajectories) is split over the ranks; N = 1 streams it in groups
# usage: tools/run_scale.sh [outdir] [extra bench.py args...]      (no curve has been measured by the builder: 1-GPU boxes only)
# NOTE strong scaling at N = 1 streams all 128 trajectories through one GPU: 60.0 s per step measured (8.1 x the shard step) -- budget
# ~5.5 min for its 2 steps + warm-up + capture + the eager profile step.
set -u
OUT=${1:-gpurun_out/scale}; shift || true
mkdir -p "$OUT"
for mode in weak strong; do
  for n in 1 2 4 8; do
    steps=4; [ "$mode" = strong ] && [ "$n" -le 2 ] && steps=2
    # the driver's verbatim form: bench.py starts its own N ranks (torch.distributed.run, one per GPU, RCCL) when N > 1
    python bench.py --gpus $n --steps $steps --warmup 1 --scaling $mode --no-cpu-baseline "$@" > "$OUT/${mode}_n$n.json" 2> "$OUT/${mode}_n$n.err"
    tail -c 400 "$OUT/${mode}_n$n.json"; echo
  done
done
python - "$OUT" <<'PY'
import glob, json, sys
for mode in ('weak', 'strong'):
    rows = {}
    for f in glob.glob(f'{sys.argv[1]}/{mode}_n*.json'):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1]); rows[d['n_gpus']] = d['value']
        except Exception:
            pass
    if 1 in rows:
        print(mode, {n: f'{v:.4f} steps/s ({v / rows[1]:.2f}x)' for n, v in sorted(rows.items())})
for f in sorted(glob.glob(f'{sys.argv[1]}/*_n*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f}: ranks seen (dist.get_world_size) {d['config'].get('ranks_seen')}, backend {d['config'].get('backend')}, "
              f"final_allgather_ms {d.get('final_allgather_ms'):.2f}, ms_per_step {d['ms_per_step']:.1f}")
    except Exception as e:
        print(f, 'unreadable:', e)
PY
