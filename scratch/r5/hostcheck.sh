cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O
(rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40) > $O/smi_idle.log
python scratch/host_ahead.py > $O/host_ahead.log 2>&1 &
PID=$!
sleep 45
(rocm-smi --showclocks --showpower 2>&1 | head -40) > $O/smi_load.log
wait $PID
tail -8 $O/host_ahead.log
grep -iE "sclk|power|mclk|fclk" $O/smi_load.log | head -12
nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null
python - <<'PY'
import time
t=time.perf_counter(); s=0
for i in range(3000000): s+=i*i
print('python loop 3e6 iters: %.3f s' % (time.perf_counter()-t))
PY
