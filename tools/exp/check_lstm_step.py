"""One-step LSTM layer kernel against a float64 torch layer: where (rows, units) the outputs differ,
for the lower layer (P = 1, 2) and the upper layer (P = 0), and a sweep of the upper layer over
sizes and seeds (three launches each).  Used when the register placement of the kernel changes."""
import sys, torch
sys.path.insert(0, ".")
from kalman_vae_amd import _capi as capi
lib = capi.load()
dev = torch.device("cuda:0")
H = 128
for P, B in ((2, 2064), (2, 4805), (2, 32768), (1, 2064), (0, 2064)):
    torch.manual_seed(B + P)
    in_dim = P if P > 0 else H
    lstm = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev)
    x = torch.randn(1, B, in_dim, device=dev)
    h_in, c_in = 0.5 * torch.randn(1, B, H, device=dev), 0.5 * torch.randn(1, B, H, device=dev)
    ref64 = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev).double()
    ref64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        _, (hn, cn) = ref64(x.double(), (h_in.double(), c_in.double()))
    w_ih = lstm.weight_ih_l0.detach()
    wx = w_ih.t().contiguous() if P == 0 else w_ih.contiguous()
    whht = lstm.weight_hh_l0.detach().t().contiguous()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
    h = torch.full((B, H), float("nan"), device=dev)
    c = torch.full((B, H), float("nan"), device=dev)
    rc = lib.kvae_lstm128_step_fused(B, P, x[0].contiguous().data_ptr(), wx.data_ptr(), h_in[0].contiguous().data_ptr(),
                                     c_in[0].contiguous().data_ptr(), whht.data_ptr(), bias.data_ptr(), h.data_ptr(), c.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    eh = (h.double() - hn[0]).abs()
    ec = (c.double() - cn[0]).abs()
    bad = (eh > 1e-4) | torch.isnan(eh)
    rows = bad.any(1).nonzero().flatten()
    units = bad.any(0).nonzero().flatten()
    print(f"P={P} B={B} rc={rc} max err h {float(eh.nan_to_num(9).max()):.2e} c {float(ec.nan_to_num(9).max()):.2e}; bad rows {rows.numel()} (first {rows[:8].tolist()}, last {rows[-4:].tolist()}), bad units {units.numel()} {units[:8].tolist()}")

# the upper layer over several sizes and seeds
for seed in range(6):
    for B in (16 * 257, 16 * 4096 + 3, 48):
        torch.manual_seed(100 + seed)
        lstm = torch.nn.LSTM(H, H, num_layers=1).to(dev)
        x = torch.randn(1, B, H, device=dev)
        h_in, c_in = 0.5 * torch.randn(1, B, H, device=dev), 0.5 * torch.randn(1, B, H, device=dev)
        ref64 = torch.nn.LSTM(H, H, num_layers=1).to(dev).double()
        ref64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
        with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
            _, (hn, cn) = ref64(x.double(), (h_in.double(), c_in.double()))
        wx = lstm.weight_ih_l0.detach().t().contiguous()
        whht = lstm.weight_hh_l0.detach().t().contiguous()
        bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
        for rep in range(3):
            h = torch.full((B, H), float("nan"), device=dev)
            c = torch.full((B, H), float("nan"), device=dev)
            lib.kvae_lstm128_step_fused(B, 0, x[0].contiguous().data_ptr(), wx.data_ptr(), h_in[0].contiguous().data_ptr(),
                                        c_in[0].contiguous().data_ptr(), whht.data_ptr(), bias.data_ptr(), h.data_ptr(),
                                        c.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            eh = float((h.double() - hn[0]).abs().nan_to_num(9).max()); ec = float((c.double() - cn[0]).abs().nan_to_num(9).max())
            if eh > 3e-5 or ec > 3e-5:
                print("UPPER LAYER BAD", seed, B, rep, eh, ec)
print("upper layer sweep done")
