# The library release whose Python surface and state_dict layout this package follows
# (reference requirements.txt:37); checkpoints carry it under 'version' (sample_ultra_res.py:56).
__version__ = '1.18.5'
