"""bench.py's parts: workload (the synthetic streams), roofline (byte models + committed counters), points (the other
operating points of the line), baselines (CPU legs + parity), multigpu (launcher, selftest, band share).  Measurement code only:
nothing here is product."""
