for rep in 1 2; do
for p in "" "-1,-1,-1,-1" "0,0,-1,-1" "0,-1,0,-1" "-1,0,0,0"; do
  if [ -z "$p" ]; then a=""; else a="--stream-priorities=$p"; fi
  v=$(python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra $a 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "priorities [$p] -> $v frames/s"
done; done
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
