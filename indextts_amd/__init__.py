"""MI355X-native IndexTTS hot-path engine (GPT speech-token decoder, s2mel flow-matching decoder, BigVGAN vocoder)."""
__version__ = "0.2.0"
