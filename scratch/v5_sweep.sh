# same-box A/B of the implicit-GEMM kernel generations / ring depths (scratch/igemm_bench.py per variant)
for cfg in "0 3 3" "1 3 3" "0 3 3" "1 3 3"; do
  set -- $cfg
  DYNMM_IGEMM_V5=$1 DYNMM_V5_SA=$2 DYNMM_V5_SB=$3 timeout 120 python scratch/igemm_bench.py 2>/dev/null | awk -v tag="V5=$1 SA=$2 SB=$3" '/^C=/{printf "%s | %s\n", tag, $0} /^sum/{printf "%s | %s\n", tag, $0}'
done
