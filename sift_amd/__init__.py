"""MI355X-native texture/noise analysis: a drop-in for the reference's
``TextureNoiseDetector`` (/root/reference/app/passive_checker.py:88-164)."""
