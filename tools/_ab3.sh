export GSDF_BENCH_MIN_WARM_S=10
for r in 1 2 3; do
for v in 0 1; do
  echo "== samples_grad_first $v run $r"
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --samples-grad-first $v 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"internal_warmup_steps": [0-9]*\|"p50": [0-9.]*' | head -3 | tr '\n' ' '; echo
done; done
