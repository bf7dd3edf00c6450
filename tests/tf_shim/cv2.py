"""TEST-ONLY: nlt/util/img.py imports cv2 at module level; nothing on the model path calls it (see README.md)."""
