"""CPU oracle for the MSSRNet hot path -- TEST INFRASTRUCTURE ONLY.

A plain-torch fp32 restatement of the reference's algorithm (functional.py: the five L1
files; phases.py: the trainer/decoder callers), pinned by the golden vectors the
reference itself produced (tests/golden/*.pt via tests/golden/make_golden.py).  It is the
checker for the HIP path and the "port" CPU baseline of bench.py; nothing in
mssrnet_style_transfer_amd/ imports it.
"""
