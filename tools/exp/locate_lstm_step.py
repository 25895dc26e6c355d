"""Where the one-step LSTM layer kernel differs from a float64 torch layer, entry by entry: for
each wrong entry the row, the unit, the tile (16 rows), the tile's place in its walker's sequence
and whether the cell state is wrong too (a wrong h with a right c is the output gate).  Every
case is launched several times on the same inputs; outputs that differ between launches are
counted apart (a deterministic miscompile gives none).  KVAE_LIB_PATH selects the build."""
import sys, torch
sys.path.insert(0, ".")
from kalman_vae_amd import _capi as capi
lib = capi.load()
dev = torch.device("cuda:0")
H = 128
CUS = torch.cuda.get_device_properties(0).multi_processor_count
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
total_bad = 0
for P, B, state in ((2, 2064, True), (2, 4805, True), (2, 32768, True), (2, 4096, True), (2, 4096, False),
                    (1, 2064, True), (1, 32768, True), (0, 2064, True), (0, 32768, True), (0, 4805, True)):
    torch.manual_seed(B + P)
    in_dim = P if P > 0 else H
    lstm = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev)
    x = torch.randn(1, B, in_dim, device=dev)
    h_in, c_in = 0.5 * torch.randn(1, B, H, device=dev), 0.5 * torch.randn(1, B, H, device=dev)
    if not state:
        h_in.zero_(); c_in.zero_()
    ref64 = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev).double()
    ref64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        _, (hn, cn) = ref64(x.double(), (h_in.double(), c_in.double()))
    w_ih = lstm.weight_ih_l0.detach()
    wx = w_ih.t().contiguous() if P == 0 else w_ih.contiguous()
    whht = lstm.weight_hh_l0.detach().t().contiguous()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
    xs, hs, cs = x[0].contiguous(), h_in[0].contiguous(), c_in[0].contiguous()
    tiles = (B + 15) // 16
    walkers = min(tiles, CUS) if P > 0 else 8 * max(1, min(CUS // 16, (tiles + 7) // 8))
    first = None
    for rep in range(REPS):
        h = torch.full((B, H), float("nan"), device=dev)
        c = torch.full((B, H), float("nan"), device=dev)
        rc = lib.kvae_lstm128_step_fused(B, P, xs.data_ptr(), wx.data_ptr(), hs.data_ptr() if state else None,
                                         cs.data_ptr() if state else None, whht.data_ptr(), bias.data_ptr(),
                                         h.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0, rc
        eh = (h.double() - hn[0]).abs()
        ec = (c.double() - cn[0]).abs()
        bad = ((eh > 1e-4) | torch.isnan(eh) | (ec > 1e-4) | torch.isnan(ec)).nonzero().tolist()
        total_bad += len(bad)
        differs = 0 if first is None else int(((h != first[0]) | (c != first[1])).sum())
        if first is None:
            first = (h.clone(), c.clone())
        print(f"P={P} B={B} state={int(state)} rep={rep}: max err h {float(eh.nan_to_num(9).max()):.2e} "
              f"c {float(ec.nan_to_num(9).max()):.2e}, wrong entries {len(bad)}, entries unlike launch 0: {differs}")
        for row, unit in bad[:24]:
            tile = row // 16
            print(f"    row {row} (tile {tile}, row {row % 16} of it; tile {tile // walkers} of walker {tile % walkers}) "
                  f"unit {unit}: h {float(h[row, unit]):+.6f} vs {float(hn[0, row, unit]):+.6f}, "
                  f"c {float(c[row, unit]):+.6f} vs {float(cn[0, row, unit]):+.6f}")
print("wrong entries in all:", total_bad)
