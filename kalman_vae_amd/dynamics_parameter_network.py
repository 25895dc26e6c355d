"""Dynamics-parameter ("weight") networks: (t,B,a_dim) -> softmax (t,B,K).

Ordinary PyTorch modules whose hot parts run on the package's HIP kernels when
the tensors are float32 on a GPU (LSTM recurrence for whole-sequence calls, the
Linear+softmax head); on the CPU they are plain torch.  Parameter names match the
reference (kvae/dynamics_parameter_network.py:18-90) so its checkpoints load:
``lstm.*`` + ``linear.*`` for the LSTM model, ``layers.<i>.*`` for the MLP.
"""
import torch
import torch.nn as nn

from . import lstm_function


class LSTMModel(nn.Module):
    """nn.LSTM -> Linear -> softmax with the recurrent state cached on the
    module between calls, so that feeding the sequence one step at a time
    equals feeding it whole."""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.lstm = nn.LSTM(input_dim, hidden_dim, num_layers=num_layers)
        self.linear = nn.Linear(hidden_dim, output_dim)
        self._hidden = None
        self.count = 0
        # whole-sequence calls on a GPU run the recurrence on the package's HIP
        # kernels (float32, hidden_dim a multiple of 32); step-by-step calls and
        # everything else use torch.nn.LSTM
        self.use_hip_recurrence = True

    @property
    def hidden(self):
        """(h_n, c_n) carried between calls, each (num_layers, B, hidden_dim), or None."""
        if isinstance(self._hidden, lstm_function.LastState):
            self._hidden = self._hidden.get()
        return self._hidden

    @hidden.setter
    def hidden(self, value):
        self._hidden = value

    def clear_hidden_state(self):
        self.hidden = None
        self.count = 0

    def reset_hidden_state(self, batch_size, device, dtype):
        shape = (self.num_layers, batch_size, self.hidden_dim)
        self.hidden = (torch.zeros(shape, dtype=dtype, device=device),
                       torch.zeros(shape, dtype=dtype, device=device))
        self.count = 0

    def forward(self, x, lead=False):
        """x (t,B,a_dim) -> weights (t,B,K); lead=True prepends the uniform weights
        of the step before the first observation: (t+1,B,K)."""
        self.count += 1
        if (self.use_hip_recurrence and self.hidden is None and x.size(0) > 1
                and lstm_function.supported(x, self.hidden_dim)):
            # whole sequence from a fresh state: recurrence on the HIP kernels (the default
            # network as one node: its backward forms the head's adjoint inside the upper
            # layer's kernel)
            fused = lstm_function.fused_lstm_head(x, self.lstm, self.linear, lead)
            if fused is not None:
                self.hidden = fused[1]
                return fused[0]
            h, hidden = lstm_function.fused_lstm(x, self.lstm)
            self.hidden = (hidden if isinstance(hidden, lstm_function.LastState)
                           else (hidden[0].detach(), hidden[1].detach()))
            return lstm_function.softmax_head(h, self.linear, lead)
        if (self.use_hip_recurrence and not torch.is_grad_enabled()
                and lstm_function.persistent_supported(x, self.lstm)):
            # inference from the carried state (the step-by-step masked filter): the
            # persistent kernel continues from self.hidden, no library RNN call
            h, self.hidden = lstm_function.lstm2_nograd(x, self.lstm, self.hidden)
            return lstm_function.softmax_head(h, self.linear, lead)
        if self.hidden is None:
            self.reset_hidden_state(x.size(1), device=x.device, dtype=x.dtype)
        h, self.hidden = self._library_lstm(x)
        return lstm_function.softmax_head(h, self.linear, lead)

    def _library_lstm(self, x):
        """self.lstm(x, self.hidden) through torch.  MIOpen refuses the backward of an LSTM
        that ran in eval mode, so a run in eval mode that records a gradient (masked
        sequences of an evaluation that differentiates) takes PyTorch's native kernels."""
        if x.is_cuda and not self.training and torch.is_grad_enabled():
            with torch.backends.cudnn.flags(enabled=False):
                return self.lstm(x, self.hidden)
        return self.lstm(x, self.hidden)

    def inference_stepper(self, x):
        """For a caller that feeds one step at a time without a gradient (the masked
        filter): an object whose ``step(x_t (B,a_dim), out (B,K))`` equals
        ``out.copy_(self(x_t[None])[0])`` with fewer launches and no per-step tensors, or
        None when that form does not apply.  Call ``finish_steps`` afterwards so that the
        module carries the state on, as after the equivalent forward calls."""
        if not self.use_hip_recurrence:
            return None
        return lstm_function.LSTM2HeadStepper.make(x, self.lstm, self.linear, self.hidden)

    def finish_steps(self, stepper, steps):
        self.hidden = stepper.hidden
        self.count += steps

    is_recurrent = True


class MLPModel(nn.Module):
    """num_layers Linear layers with ReLU between them, softmax on top."""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.input_dim = input_dim
        self.output_dim = output_dim
        # The first layer always maps to hidden_dim, also when it is the only
        # layer (dynamics_parameter_network.py:70-76 tests i == 0 first).
        self.layers = nn.ModuleList()
        for i in range(num_layers):
            fan_in = input_dim if i == 0 else hidden_dim
            fan_out = output_dim if (i == num_layers - 1 and i != 0) else hidden_dim
            self.layers.append(nn.Linear(fan_in, fan_out))

    def clear_hidden_state(self):
        pass

    def reset_hidden_state(self, batch_size, device, dtype):
        pass

    def forward(self, x, lead=False):
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            if i == last:
                return lstm_function.softmax_head(x, layer, lead)
            x = torch.relu(lstm_function.linear(x, layer))

    is_recurrent = False


def make_weight_model(kind, a_dim, K, hidden_dim, num_layers):
    if kind == "mlp":
        return MLPModel(a_dim, K, hidden_dim=hidden_dim, num_layers=num_layers)
    if kind == "lstm":
        return LSTMModel(a_dim, K, hidden_dim=hidden_dim, num_layers=num_layers)
    raise ValueError("Invalid dynamics_parameter_network: {}".format(kind))
