"""Empty stub for `seaborn` (absent in this image)."""
