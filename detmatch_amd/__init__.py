"""detmatch_amd — MI355X-native DetMatch training step."""
