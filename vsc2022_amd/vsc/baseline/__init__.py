"""Mirror of `vsc.baseline` (score normalisation, localisation, the SSCD matching driver)."""
