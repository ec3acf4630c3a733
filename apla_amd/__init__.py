"""apla_amd — MI355X-native APLA fine-tuning hot path.

Importing the package does not touch the GPU or load the HIP library; kernels are bound lazily (apla_amd._lib.lib()).
"""
__version__ = "0.1.0"
