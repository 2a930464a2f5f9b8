"""Python adaptors (host-side mirror of envpool/python/ of the reference)."""
