"""Empty stand-in: the reference imports pygame at module level but only uses it for render_mode="human"."""
