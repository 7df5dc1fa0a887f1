"""MI355X-native IndexTTS hot-path engine (GPT speech-token decoder + BigVGAN vocoder).

Import name: `indextts_amd` (the directory name `index-tts_amd` is not a valid identifier; the sibling
`indextts_amd/` package maps the name onto this directory).
"""
__version__ = "0.1.0"
