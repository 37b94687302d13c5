cd $GRAFT_REPO_ROOT
for o in "" "2=0" "7=0" "8=0" "5=192" "5=224"; do
  echo "opts [$o]: $(MEDMOE_OPTS=$o python bench.py --steps 20 --warmup 5 --global-batch 128 --no-cpu-baseline --path engine 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
python bench.py --config ref_swin --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-400
