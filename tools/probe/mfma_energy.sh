# on the GPU box:  bash tools/probe/mfma_energy.sh   (tools/probe/bin/mfma_energy built by: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_energy.cpp)
cd $GRAFT_REPO_ROOT
for zero in 0 1; do for mode in 0 1 2; do
  ./tools/probe/bin/mfma_energy $mode $zero 4 &
  pid=$!
  sleep 2.5
  rocm-smi --showpower --showclocks --json | python3 -c "import json,sys; d=json.load(sys.stdin); c=d.get('card0', list(d.values())[0]); print('   ', {k: v for k, v in c.items() if 'ower' in k or 'sclk clock speed' in k})"
  wait $pid
done; done
