for t in 64 128; do
echo "== DYNMM_IGEMM_TPIX=$t"
DYNMM_IGEMM_TPIX=$t python scratch/ksweep.py 2>&1 | grep "KH=3"
DYNMM_IGEMM_TPIX=$t python scratch/trace/run_trace.py 2>&1 | grep -A4 "^C=128 60x80 k3x1"
done
