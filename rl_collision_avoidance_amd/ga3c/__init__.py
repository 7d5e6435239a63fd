"""Host-side mirrors of the GA3C pieces on the hot path (ga3c/GA3C/Environment.py,
ProcessAgent.py, Experience.py of the reference)."""
