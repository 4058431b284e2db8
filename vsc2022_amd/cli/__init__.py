"""Command-line drivers with the flags of the reference's descriptor_eval.py / matching_eval.py /
`python -m vsc.baseline.sscd_baseline`, running on the MI355X engine."""
