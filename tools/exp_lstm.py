"""Experiment: cost of the PyTorch LSTM weight network alone (fwd+bwd) at the
benchmark shape, MIOpen vs PyTorch-native kernels."""
import time, torch
from kalman_vae_amd.dynamics_parameter_network import LSTMModel, MLPModel
dev = torch.device("cuda:0")
T, B = 20, 4096
for name, ctx in (("miopen", True), ("native", False)):
    with torch.backends.cudnn.flags(enabled=ctx):
        net = LSTMModel(2, 3, 128, 2).to(dev).train()
        a = torch.randn(T, B, 2, device=dev, requires_grad=True)
        def step():
            net.zero_grad(set_to_none=True); a.grad = None
            net.clear_hidden_state()
            w = net(a)
            (w * w).sum().backward()
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); print(name, "LSTM fwd+bwd ms", (time.perf_counter() - t0) / 20 * 1e3)
        def fwd():
            with torch.no_grad():
                net.clear_hidden_state(); net(a)
        for _ in range(5): fwd()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fwd()
        torch.cuda.synchronize(); print(name, "LSTM fwd ms", (time.perf_counter() - t0) / 20 * 1e3)
net = MLPModel(2, 3, 128, 2).to(dev).train()
a = torch.randn(T, B, 2, device=dev, requires_grad=True)
def step():
    net.zero_grad(set_to_none=True); a.grad = None
    w = net(a); (w * w).sum().backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("MLP fwd+bwd ms", (time.perf_counter() - t0) / 20 * 1e3)
