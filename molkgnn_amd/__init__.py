"""MI355X-native molecular-kernel convolution (MolKGNN hot path)."""
__version__ = "0.1.0"
