"""Drop-in for the names the reference imports from `imagen_pytorch`
(train_ultra_res.py:8, sample_ultra_res.py:12-13), backed by the MI355X HIP engine."""
from .imagen_pytorch import ElucidatedImagen, Imagen, NullUnet, SRUnet1024, Unet
from .trainer import ImagenTrainer, restore_parts
from .version import __version__

__all__ = ["Unet", "Imagen", "NullUnet", "SRUnet1024", "ElucidatedImagen", "ImagenTrainer", "restore_parts",
           "__version__"]
