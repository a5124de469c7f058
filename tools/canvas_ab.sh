cd $GRAFT_REPO_ROOT
for o in 1 2 1 2; do
  echo "== bench RN_CANVAS_SLOTS=$o"; RN_CANVAS_SLOTS=$o python bench.py --no-detect --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['conv_mfma'])"
done
