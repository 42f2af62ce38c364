"""NB-BLSTM — drop-in for the reference's models/arch/blstm2_fc1.py (same constructor :8-16, forward [B,F,T,dim_input] ->
[B,F,T,dim_output] :45-68, same state_dict keys `blstm1.* / blstm2.* / linear.*`).  One BiLSTM stack per frequency bin
(the F axis is folded into the batch), two recurrent layers and one linear map.  Plain PyTorch on the CPU (BASELINE
config 1 is a CPU plumbing run); on a HIP device the native path of nbss_amd/blstm.py — one persistent launch per layer for the recurrences of both
directions, dense maps and contractions around it — for inference and training (hidden sizes 128 / 256: the shipped configuration)."""
import os
import warnings
import weakref
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

_NATIVE = weakref.WeakKeyDictionary()  # module -> (nbss_amd.blstm.NativeBLSTM or None, reason it is None)
_NOTED = weakref.WeakKeyDictionary()   # module -> reasons already reported


class BLSTM2_FC1(nn.Module):
    def __init__(self, dim_input: int, dim_output: int, activation: Optional[str] = "", hidden_size: Tuple[int, int] = (256, 128),
                 n_repeat_last_lstm: int = 1, dropout: Optional[float] = None):
        super().__init__()
        h1, h2 = hidden_size
        self.input_size, self.output_size, self.hidden_size = dim_input, dim_output, tuple(hidden_size)
        self.activation, self.dropout = activation, dropout
        self.blstm1 = nn.LSTM(dim_input, h1, batch_first=True, bidirectional=True)
        self.blstm2 = nn.LSTM(2 * h1, h2, num_layers=n_repeat_last_lstm, batch_first=True, bidirectional=True)
        if dropout is not None:
            self.dropout1, self.dropout2 = nn.Dropout(dropout), nn.Dropout(dropout)
        self.linear = nn.Linear(2 * h2, dim_output)
        self.activation_func = getattr(nn, activation)() if activation else None

    def _native(self):
        if self not in _NATIVE:
            runner, why = None, None
            try:
                from nbss_amd._lib import hip
                from nbss_amd.blstm import NativeBLSTM, supported
                why = supported(self)
                if why is None:
                    runner = NativeBLSTM(self, hip())
            except Exception as e:  # (no library / no HIP runtime: torch.nn below)
                runner, why = None, f"{type(e).__name__}: {e}"
            _NATIVE[self] = (runner, why)
        return _NATIVE[self][0]

    def forward(self, x: Tensor) -> Tensor:
        B, F, T, _ = x.shape
        if x.is_cuda:  # the native path; otherwise (one warning per reason) the torch.nn modules below
            why = None
            if os.environ.get("NBSS_BLSTM_NATIVE", "1") == "0":
                why = "NBSS_BLSTM_NATIVE=0"
            elif x.dtype not in (torch.float32, torch.bfloat16):
                why = f"input dtype {x.dtype}"
            elif any(p.dtype != torch.float32 for p in self.parameters()):
                why = "parameters are not fp32"
            elif self._native() is None:
                why = _NATIVE[self][1] or "native path unavailable"
            elif torch.is_grad_enabled() and x.requires_grad:
                why = "the input requires a gradient (the native backward produces parameter gradients only)"
            if why is None:
                return self._native().forward_train(x.contiguous()) if torch.is_grad_enabled() else self._native().forward(x.contiguous())
            seen = _NOTED.setdefault(self, set())
            if why not in seen:
                seen.add(why)
                warnings.warn(f"NB-BLSTM: torch.nn path instead of the native HIP kernels ({why})", RuntimeWarning, stacklevel=2)
        h = x.reshape(B * F, T, -1)  # every frequency is an independent sequence
        for rnn, drop in ((self.blstm1, "dropout1"), (self.blstm2, "dropout2")):
            h, _ = rnn(h)
            if self.dropout:
                h = getattr(self, drop)(h)
        y = self.linear(h)
        if self.activation_func is not None:
            y = self.activation_func(y)
        return y.reshape(B, F, T, -1)
