import os, sys, torch
sys.path.insert(0, "/root/repo")
from kalman_vae_amd import _capi as capi
if os.environ.get("KVAE_ALT_LIB"):
    capi.LIB_PATH = os.environ["KVAE_ALT_LIB"]
dev = torch.device("cuda:0"); T, P, H, B = 20, 2, 128, 4096
lib = capi.load()
x = torch.randn(T, B, P, device=dev)
w = lambda *s: torch.randn(*s, device=dev) * 0.08
wih0, b0, whh0T, wih1T, whh1T, b1 = w(4*H, P), w(4*H), w(H, 4*H), w(H, 4*H), w(H, 4*H), w(4*H)
new = lambda *s: torch.empty(*s, device=dev)
g0, h0, c0, g1, h1, c1 = new(T,B,4*H), new(T,B,H), new(T,B,H), new(T,B,4*H), new(T,B,H), new(T,B,H)
st = capi.current_stream_ptr()
def run(gates):
    a, b = (g0.data_ptr(), g1.data_ptr()) if gates else (None, None)
    capi.check(lib.kvae_lstm2_forward(T,B,P,x.data_ptr(),wih0.data_ptr(),b0.data_ptr(),whh0T.data_ptr(),wih1T.data_ptr(),whh1T.data_ptr(),b1.data_ptr(),a,h0.data_ptr(),c0.data_ptr(),b,h1.data_ptr(),c1.data_ptr(),None,None,st))
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
print(os.environ.get("KVAE_ALT_LIB","default"), "fwd with gates %.1f us, without gate stores %.1f us" % (timed(lambda: run(True)), timed(lambda: run(False))))
