"""Mirror of the reference's ``TextMelCollateFn`` (models/tts/__init__.py:28-62): the batch layout ``Tacotron2.forward`` consumes.

Rows are sorted by token count, longest first (the reference's packed BiLSTM requires it and the kernels keep that
contract); tokens are zero padded, mels are zero padded along time, the gate target is 1 from each row's last frame on."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch


class TextMelCollateFn:
    def __call__(self, batch: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        # same expression as the reference (models/tts/__init__.py:32) so that rows with equal token counts come out in the
        # reference's order as well
        order = [int(i) for i in np.argsort([x["tokens"].shape[0] for x in batch])[::-1]]
        n_mels = batch[0]["features"].shape[0]
        L = batch[order[0]]["tokens"].shape[0]
        T = max(x["features"].shape[1] for x in batch)
        B = len(batch)
        out = {"token_padded": torch.zeros(B, L, dtype=torch.long), "token_lengths": torch.zeros(B, dtype=torch.long),
               "mel_padded": torch.zeros(B, n_mels, T), "gate_padded": torch.zeros(B, T), "mel_lengths": torch.zeros(B, dtype=torch.long)}
        for row, i in enumerate(order):
            tok, mel = batch[i]["tokens"], batch[i]["features"]
            out["token_padded"][row, : tok.shape[0]] = tok
            out["token_lengths"][row] = tok.shape[0]
            out["mel_padded"][row, :, : mel.shape[1]] = mel
            out["gate_padded"][row, mel.shape[1] - 1:] = 1
            out["mel_lengths"][row] = mel.shape[1]
        return out
