cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6u; rm -rf $O; mkdir -p $O
cat > /tmp/conv_micro4.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from dynmm_amd import lib as L
if sys.argv[6] != '-':
    L.LIB_PATH = os.path.join(os.environ['GRAFT_REPO_ROOT'], sys.argv[6])
import torch
from dynmm_amd import ops
N, C, H, W, KH, KW = 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
x = torch.randn(N, C, H, W, device='cuda', requires_grad=True)
w = (torch.randn(C, C, KH, KW, device='cuda') * 0.05).requires_grad_(True)
b = torch.zeros(C, device='cuda', requires_grad=True)
g = torch.randn(N, C, H, W, device='cuda')
for _ in range(3):
    y = ops.conv2d(x, w, b, 1, (KH // 2, KW // 2), None)
    y.backward(g)
torch.cuda.synchronize()
PY
for shape in "128 60 80" "256 30 40"; do
for lib in - scratch/r6/libdynmm_vtskew1.so scratch/r6/libdynmm_vtskew2.so; do
  tag=$(echo "$shape $(basename $lib .so)" | tr ' ' '_')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$tag -o p -- python /tmp/conv_micro4.py $shape 3 1 $lib > $O/$tag.log 2>&1
  python - <<PY
import csv, collections, glob
f = glob.glob('$O/$tag/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'wgrad_wino_vt' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:50]].append(float(r['Counter_Value']))
C, H, W = [int(v) for v in "$shape".split()]
alg = 2 * 32 * C * H * W * 4 / 1048576.0
for k, v in agg.items():
    m = sum(v) / len(v) / 1024
    print("$tag", 'raw MiB', round(m, 1), 'ratio(x2)', round(2 * m / alg, 2))
PY
  rm -rf $O/$tag
done
done
