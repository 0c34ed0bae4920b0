"""Developer probe: the classifier GEMM at the decode shape (1280 rows x 36541 tokens x 512) with and without the logits
store (dh_vocab_logits vs dh_vocab_logprob)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
hip.load()
dev = "cuda"
M, V, K = int(os.environ.get("KB_M", 1280)), int(os.environ.get("KB_V", 36541)), int(os.environ.get("KB_K", 512))
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("KB_DTYPE", "bf16")]
a = [(torch.randn(M, K, device=dev)).to(DT) for _ in range(4)]
w = (torch.randn(V, K, device=dev) * K ** -0.5).to(DT)
b = torch.zeros(V, device=dev)
ldl = (V + 127) // 128 * 128                       # whole 128-column panels, as the decoders allocate it
logits = torch.empty(M, ldl, device=dev)[:, :V]
gm = torch.empty(M, hip.n_groups(V), device=dev)
tg = torch.randint(0, V, (M,), device=dev)
def timeit(fn, iters=40, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    with hip.profile() as prof:
        for i in range(iters): fn(i)
        torch.cuda.synchronize()
    return {k: round(v["ms"] / v["calls"] * 1e3, 1) for k, v in prof.summary().items()}
print("logits ", timeit(lambda i: hip.vocab_logits(a[i % 4], w, b, logits, gm)))
print("gmax   ", timeit(lambda i: hip.vocab_logits(a[i % 4], w, b, None, gm)))
print("logprob", timeit(lambda i: hip.vocab_logprob(a[i % 4], w, b, tg)))
vpad = (V + 255) // 256 * 256
if K == 512 and hip.vocab_logits_wreg_supported(M, V, K, vpad, vpad // 64):      # round 4: weights streamed from L2 into registers
    wp, bp = hip.pack_vocab_weights(w, b)
    lg2, gm2 = torch.empty(M, vpad, device=dev), torch.empty(M, vpad // 64, device=dev)
    print("wreg logits ", timeit(lambda i: hip.vocab_logits_wreg(a[i % 4], wp, bp, V, lg2, gm2)))
    print("wreg gmax   ", timeit(lambda i: hip.vocab_logits_wreg(a[i % 4], wp, bp, V, None, gm2)))
