"""arcle_amd — MI355X-native implementation of ARCLE's data-parallel hot path."""
__version__ = "0.1.0"
