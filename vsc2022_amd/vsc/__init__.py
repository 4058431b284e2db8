"""Host-side mirror of the reference's `vsc` package (hot path only)."""
