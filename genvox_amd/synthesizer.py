"""Mirror of the reference's ``core.synthesizer.Synthesizer`` (core/synthesizer.py:9-45): config + checkpoint
loading, text -> tokens, autoregressive Tacotron2 on the GPU, mel -> waveform on the GPU."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .audio import AudioProcessor
from .text import TextProcessor


class Synthesizer:
    def __init__(self, tts_model_class, tts_config_path: str, tts_checkpoint_path: str, use_cuda: bool = True) -> None:
        if not (use_cuda and torch.cuda.is_available()):
            raise RuntimeError("genvox_amd.Synthesizer needs an MI355X (use_cuda=True and a visible GPU); it has no CPU path")
        self.device = "cuda:0"
        self.tts_model = tts_model_class.load_from_config(config_path=tts_config_path)
        self.tts_model.to(self.device)
        self.tts_model.eval()
        ckpt = torch.load(tts_checkpoint_path, map_location="cpu")
        print(f"[genvox_amd] {self.tts_model.model_name} on {self.device} (MI355X HIP path); "
              f"checkpoint {tts_checkpoint_path} at iteration {ckpt.get('iteration')}")
        self.tts_model.load_checkpoint_statedicts(statedicts=ckpt, save_optimizer_dict=False, optimizer=None)
        self.text_processor = TextProcessor(config=self.tts_model.text_config)
        self.audio_processor = AudioProcessor(config=self.tts_model.audio_config, device=self.device)

    def tts(self, text: str) -> Dict[str, np.ndarray]:
        tokens = self.text_processor.tokens_to_indices(self.text_processor.tokenize(text))
        tokens = torch.IntTensor(tokens).unsqueeze(0).to(self.device)
        outputs = self.tts_model.inference(inputs={"tokens": tokens})
        mel = outputs["mel_outputs_postnet"]
        wav = self.audio_processor.convert_mel2wav_batch(mel)  # stays on the device until the end
        result = {key: val.squeeze(0).cpu().numpy() for key, val in outputs.items()}
        result["waveform"] = wav[0].cpu().numpy()
        result["sampling_rate"] = self.audio_processor.config.sampling_rate
        return result
