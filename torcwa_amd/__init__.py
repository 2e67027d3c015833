"""torcwa_amd: MI355X-native RCWA inner solver, drop-in for the torcwa.rcwa hot path."""
__version__ = "0.1.0"
