"""Real photographs for the tests: scikit-learn ships two (china.jpg, flower.jpg, 427 x 640) inside its installed
package; they are read from there at run time (nothing of them is copied into this repository) and put through the
quality ladder the authors of SSIMULACRA2 published as the metric's calibration:

    score  10  very low quality   libjpeg-turbo quality 14, 4:2:0
           30  low quality        libjpeg-turbo quality 20, 4:2:0
           50  medium quality     libjpeg-turbo quality 35, 4:2:0
           70  high quality       libjpeg-turbo quality 70, 4:2:0
           80  very high quality  libjpeg-turbo quality 85, 4:2:2
           85  excellent quality  libjpeg-turbo quality 90, 4:4:4
           90  visually lossless  libjpeg-turbo quality 95, 4:4:4

(the README of SSIMULACRA 2, "average output of" each setting over the authors' corpus; the table is not on this
disk and is written here from the published text, so this is a WEAK external pin -- but an external one: nothing in
it comes from this repository).  Pillow's JPEG codec is libjpeg-turbo."""
import io

import numpy as np

LADDER = [(14, 2, 10.0), (20, 2, 30.0), (35, 2, 50.0), (70, 2, 70.0), (85, 1, 80.0), (90, 0, 85.0), (95, 0, 90.0)]
#          quality, Pillow subsampling code (2 = 4:2:0, 1 = 4:2:2, 0 = 4:4:4), published score


def photographs():
    """-> [(name, (h, w, 3) uint8)] or [] when scikit-learn's sample images are not installed"""
    try:
        from sklearn.datasets import load_sample_images
        ds = load_sample_images()
    except Exception:  # noqa: BLE001  (missing package, missing files)
        return []
    return [(str(n).split("/")[-1], np.ascontiguousarray(np.asarray(im)[..., :3])) for n, im in zip(ds.filenames, ds.images)]


def jpeg_round_trip(rgb: np.ndarray, quality: int, subsampling: int) -> np.ndarray:
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, format="JPEG", quality=int(quality), subsampling=int(subsampling))
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB")))
