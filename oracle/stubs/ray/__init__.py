"""Minimal stand-in for `ray` so that the reference's research env (agents/env.py) and the policy-side observation
layout helper (agents/models/GPTPolicy.py: unflatten_vec) can be imported in the build container.  Test infrastructure
only (tests/golden/make_golden_research.py); no arithmetic of the hot path lives here."""
