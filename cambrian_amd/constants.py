"""cambrian/constants.py:7-8 — the two constants that select the hot-path branch."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
# cambrian/constants.py:9-13 — special tokens the loaders add to the tokenizer
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
