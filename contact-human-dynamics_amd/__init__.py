"""MI355X-native physics hot path of contact-human-dynamics."""
