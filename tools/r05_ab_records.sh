tools/ab_bench.sh 4 tree ab_variants/lib_before_records.so
echo "--- C5"
for r in 1 2 3; do for lib in tree ab_variants/lib_before_records.so; do
 if [ $lib = tree ]; then v=$(python bench.py --splats 5000000 --width 3840 --height 2160 --steps 64 --warmup 8 --cpu-frames 0 --no-extra | tail -1 | grep -o '"value": [0-9.]*'); else v=$(GSPLAT_HIP_LIBRARY=$PWD/$lib python bench.py --splats 5000000 --width 3840 --height 2160 --steps 64 --warmup 8 --cpu-frames 0 --no-extra | tail -1 | grep -o '"value": [0-9.]*'); fi; echo "C5 $lib $v"; done; done
