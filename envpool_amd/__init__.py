"""envpool_amd: MI355X-native batched-step engine behind envpool's API."""
__version__ = "0.1.0"
