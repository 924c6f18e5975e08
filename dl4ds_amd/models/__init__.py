from .sp_postups import net_postupsampling
from .sp_preups import net_pin, unet_pin
from .spt_postups import recnet_postupsampling
from .spt_preups import recnet_pin
from .discriminator import residual_discriminator
