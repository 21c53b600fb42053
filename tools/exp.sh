cd $GRAFT_REPO_ROOT
L=nndetection_amd/csrc/libnndet_amd.so
cp $L /tmp/orig.so
for v in orig a b c d orig; do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp build/lib$v.so $L; fi
  echo "== $v $(MICRO_ORDER=fwd,dgrad,fwd,dgrad MICRO_ITERS=40 timeout 300 python tools/conv_microbench.py e0_32x32_full 2>&1 | grep -v "Warn\|amdgpu" | sed 's/.*MB |//; s/GB\/s//g; s/[0-9.]* TF\/s *[0-9]*//g')"
done
cp /tmp/orig.so $L
