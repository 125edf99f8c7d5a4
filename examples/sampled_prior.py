"""Kept for callers of round 4's layout: the sampled-prior funnel lives in the package now
(`sober_amd/_sampled_prior.py`) and is `sober_amd.Sober`'s default `candidate_funnel`."""
from sober_amd._sampled_prior import sampling_candidates  # noqa: F401
