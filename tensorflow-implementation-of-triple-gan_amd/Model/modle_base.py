"""The reference's file is spelled Model/modle_base.py while its models import `model_base`
(Model/Good_GAN.py:8); both names resolve to the same module here."""
from Model.model_base import *          # noqa: F401,F403
from Model.model_base import NN_Base    # noqa: F401
