"""Mirror of the `vcsl` package surface the reference imports (`vcsl.vta.build_vta_model`)."""
