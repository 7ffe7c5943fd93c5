# bench.py --gpus 8 with the eight ranks sharing the one GPU over gloo, repeatedly; the first failing run's stderr -> gpurun_out/eight_fail.err
cd $GRAFT_REPO_ROOT
free -g | head -2; nproc
for i in 1 2 3 4 5 6; do
  for c in cfg2 cfg5; do
    t0=$(date +%s); LPM_SHARE_GPU=1 timeout 600 python bench.py --config $c --gpus 8 --steps 2 --warmup 1 --spinup-seconds 0 --no-cpu-baseline > /tmp/o.log 2> /tmp/e.log
    rc=$?; echo "run $i $c rc=$rc $(( $(date +%s) - t0 )) s"
    if [ $rc != 0 ]; then grep -v "hostname of the client\|amdgpu.ids" /tmp/e.log > gpurun_out/eight_fail.err; dmesg 2>/dev/null | tail -5; exit 0; fi
  done
done
