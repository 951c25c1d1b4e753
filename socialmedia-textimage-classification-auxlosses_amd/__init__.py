"""mmhip: MI355X-native late-fusion fine-tuning path."""
