python -m pytest tests/test_hip_model.py -m gpu -q --timeout=900 -k "oracle_fwd_bwd" 2>&1 | tail -3
python - <<'PY'
import sys; sys.path.insert(0,'.')
import torch
from dynmm_amd import synth, ops, dp
from dynmm_amd.nn.net import SkipGateESANet
# determinism / race screen: dual-stream eager vs single-stream, full-size batch 8, 3 repeats, with DIRECT_GRAD
def run(dual, reps=3):
    torch.manual_seed(0)
    m = SkipGateESANet(encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), 0)
    m = m.cuda().train(); m.dual_stream = dual
    red = dp.GradBucketReducer(m.parameters(), overlap=False); ops.DIRECT_GRAD = True
    rgb, depth = synth.synth_inputs(8, 480, 640, seed=5, device='cuda')
    outs_all = []
    for _ in range(reps):
        synth.fill_state_dict(m.state_dict(), 0)
        red.zero()
        outs, lf = m(rgb, depth)
        (sum(o.square().mean() for o in outs) + lf).backward()
        torch.cuda.synchronize()
        outs_all.append((outs[0].detach().clone(), red.flat.clone()))
    return outs_all
a = run(False); b = run(True)
for i in range(3):
    print('rep', i, 'logits max|diff|', (a[0][0]-b[i][0]).abs().max().item(), 'grad rel diff', ((a[0][1]-b[i][1]).norm()/a[0][1].norm()).item(),
          'single-stream repeatability', ((a[0][1]-a[i][1]).norm()/a[0][1].norm()).item())
PY
