"""TEST INFRASTRUCTURE (CPU oracle, see oracle/__init__.py): the reference's per-pixel latent-vector export, restated verbatim
in behaviour: /root/reference/pixel_latent_vector.py:48-55 (`generate_title`) and :85-101 (per-pixel dict in row-major pixel
order, key = the (row, col) tuple, values = the N grey levels followed by the label, written with csv.writer into a file opened
with newline='').  Pure-Python loops: use on small images only."""
import csv
import io

import numpy as np


def generate_title(n):
    return ["Pixel No."] + ["Sample " + str(i + 1) for i in range(n)] + ["Category"]


def pixel_csv_bytes(grayscale_images, label_hw) -> bytes:
    """grayscale_images: list of N uint8 [H, W] arrays; label_hw: [H, W] array (the reference indexes label[0][0])."""
    n = len(grayscale_images)
    height, width = grayscale_images[0].shape
    pixel_dict = {}
    for i in range(height):
        for j in range(width):
            vec = [grayscale_images[k][i, j] for k in range(n)]
            vec.append(label_hw[i, j])
            pixel_dict[(i, j)] = vec
    buf = io.StringIO(newline="")
    writer = csv.writer(buf)
    writer.writerow(generate_title(n))
    for key, values in pixel_dict.items():
        writer.writerow([key] + values)
    return buf.getvalue().encode()
