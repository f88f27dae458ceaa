"""MI355X-native radiosonde IQ demodulation engine (hot path of radiosonde_auto_rx)."""
__version__ = "0.1.0"
