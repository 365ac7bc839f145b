"""CPU oracles (test infrastructure only): see oracle/progan.py and oracle/audio.py headers."""
