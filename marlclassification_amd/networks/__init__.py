from .models import ModelOutput, ModelsWrapper, RecurrentOutput
from .vision import VisionCnnModule
