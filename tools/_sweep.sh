cd $GRAFT_REPO_ROOT
for v in 2048 3200 4224 6400 8448 12672; do
  echo "min_rows $v: $(MEDMOE_OPTS=9=$v python bench.py --steps 20 --warmup 5 --global-batch 128 --no-cpu-baseline --path engine 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)  serial: $(MEDMOE_OVERLAP_WGRAD=0 MEDMOE_OPTS=9=$v python bench.py --steps 20 --warmup 5 --global-batch 128 --no-cpu-baseline --path engine 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
