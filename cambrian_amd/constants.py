"""cambrian/constants.py:7-8 — the two constants that select the hot-path branch."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
