"""genvox_amd: MI355X-native Tacotron2 text->mel forward path and Griffin-Lim vocoder behind GenVox's Python surface.

    from genvox_amd import Tacotron2, Synthesizer, AudioProcessor, Tacotron2Config, AudioConfig, TextConfig
"""
from .configs import AudioConfig, BaseConfig, Tacotron2Config, TextConfig  # noqa: F401


def __getattr__(name):  # torch-dependent classes are imported lazily
    if name == "Tacotron2":
        from .tacotron2 import Tacotron2
        return Tacotron2
    if name == "Synthesizer":
        from .synthesizer import Synthesizer
        return Synthesizer
    if name == "AudioProcessor":
        from .audio import AudioProcessor
        return AudioProcessor
    if name == "TextProcessor":
        from .text import TextProcessor
        return TextProcessor
    raise AttributeError(name)
