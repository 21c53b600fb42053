cd $GRAFT_REPO_ROOT
L=nndetection_amd/csrc/libnndet_amd.so
cp $L /tmp/new.so
# the old library has the 3-kernel norm backward and needs the old red_ws layout: it ignores the extra N doubles -> compatible
for v in new old nofence new old nofence; do
  cp build/lib$v.so $L 2>/dev/null || cp /tmp/new.so $L
  echo -n "$v: "; python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"
done
cp /tmp/new.so $L
