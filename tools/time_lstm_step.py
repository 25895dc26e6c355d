"""One-step (T=1) calls of the persistent LSTM kernel from a carried state, as the masked
filter makes them: both tile heights.  usage: python tools/time_lstm_step.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalman_vae_amd import lstm_function as LF  # noqa: E402

dev = torch.device("cuda:0")
lstm = torch.nn.LSTM(2, 128, num_layers=2).to(dev)
for B in [int(a) for a in sys.argv[1:]] or [65536]:
    x = torch.randn(1, B, 2, device=dev)
    hidden = (torch.randn(2, B, 128, device=dev), torch.randn(2, B, 128, device=dev))
    for rows in (16, 32):
        os.environ["KVAE_LSTM_TILE_ROWS"] = str(rows)
        for _ in range(3):
            LF.lstm2_nograd(x, lstm, hidden)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            LF.lstm2_nograd(x, lstm, hidden)
        e1.record()
        torch.cuda.synchronize()
        print(f"B={B} T=1 rows={rows}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per call (incl. operand preparation)")
    os.environ.pop("KVAE_LSTM_TILE_ROWS")
