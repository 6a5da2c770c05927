"""aggfly_amd — MI355X-native engine for aggfly's aggregate_dataset() hot path."""
__version__ = "0.1.0"
