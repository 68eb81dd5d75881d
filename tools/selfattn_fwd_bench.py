import sys, torch, json
from sta import ops
def timed(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for B, N, C, h in ((64, 4096, 320, 8), (64, 1024, 640, 8)):
    d = C // h
    qk = torch.randn(B, N, 2 * C, device="cuda", dtype=torch.float16)
    vt = torch.randn(B, C, N, device="cuda", dtype=torch.float16)
    a = timed(lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, d ** -0.5))
    b = timed(lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, ops.LN2))
    print(json.dumps({"B": B, "N": N, "d": d, "scaled_us": round(a, 1), "log2_us": round(b, 1)}))
