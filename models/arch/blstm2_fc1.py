"""NB-BLSTM — drop-in for the reference's models/arch/blstm2_fc1.py (same constructor :8-16, forward [B,F,T,dim_input] ->
[B,F,T,dim_output] :45-68, same state_dict keys `blstm1.* / blstm2.* / linear.*`).  One BiLSTM stack per frequency bin
(the F axis is folded into the batch), two recurrent layers and one linear map.  Plain PyTorch: SURVEY.md §8(f) rank 3 (a
BASELINE config-1 plumbing model; the recurrences run in torch's LSTM on whatever device the module lives on)."""
from typing import Optional, Tuple

import torch.nn as nn
from torch import Tensor


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

    def forward(self, x: Tensor) -> Tensor:
        B, F, T, _ = x.shape
        h = x.reshape(B * F, T, -1)  # every frequency is an independent sequence
        for rnn, drop in ((self.blstm1, "dropout1"), (self.blstm2, "dropout2")):
            h, _ = rnn(h)
            if self.dropout:
                h = getattr(self, drop)(h)
        y = self.linear(h)
        if self.activation_func is not None:
            y = self.activation_func(y)
        return y.reshape(B, F, T, -1)
