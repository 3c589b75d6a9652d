"""wseg_amd — MI355X-native (gfx950) implementation of the wseg contrast_train hot path."""
__version__ = "0.1.0"
