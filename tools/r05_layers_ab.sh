for rep in 1 2; do
for a in "" "--depth-layers 2" "--depth-layers 3" "--depth-layers 4" "--depth-layers 2 --layer-fractions 0.1" "--depth-layers 2 --layer-fractions 0.25" "--depth-layers 3 --layer-fractions 0.1,0.3"; do
  v=$(python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['extra']['latency']['one_frame_then_sync']['ms'] if d.get('extra') else '')")
  echo "[$a] -> $v"
done; done
