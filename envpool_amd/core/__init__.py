"""Binding layer between the Python adaptors and the C ABI (include/envpool_amd.h)."""
