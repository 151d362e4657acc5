"""Image transport helpers for vision-language rollouts (``areal/utils/image.py:1-43``)."""
import base64
from io import BytesIO
from typing import List, Union


def image2base64(images: Union[List, object]) -> List[str]:
    """PIL images or encoded bytes -> base64 PNG strings (the wire format of image requests)."""
    if not isinstance(images, (list, tuple)):
        images = [images]
    out = []
    for im in images:
        if isinstance(im, (bytes, bytearray)):
            out.append(base64.b64encode(bytes(im)).decode("utf-8"))
            continue
        buf = BytesIO()
        im.save(buf, format="PNG")
        out.append(base64.b64encode(buf.getvalue()).decode("utf-8"))
    return out


def base642image(data: str):
    from PIL import Image

    return Image.open(BytesIO(base64.b64decode(data)))


def pad_images_batch_to_max_size(images: List):
    """Pad PIL images (top-left anchored, black fill) to the largest width/height of the batch."""
    from PIL import Image

    mw, mh = max(i.width for i in images), max(i.height for i in images)
    out = []
    for im in images:
        canvas = Image.new("RGB", (mw, mh), (0, 0, 0))
        canvas.paste(im, (0, 0))
        out.append(canvas)
    return out
