"""cosmogp_amd: MI355X-native GP hot path."""
