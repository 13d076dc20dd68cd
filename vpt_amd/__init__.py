"""vpt_amd — MI355X-native re-host of the MIP / EAM / MCS / MCM renderer path of MOj0/vpt.

Host-side mirror of the reference's Renderer plugin surface over a C-ABI HIP library
(include/vpt.h -> vpt_amd/libvpt_hip.so).  Importing the package does not load the library;
constructing a Context does, and raises if it is missing (no CPU fallback)."""
from .property_bag import PropertyBag, EventTarget, Event, CustomEvent
from .scene import Node, Transform, PerspectiveCamera, mat4, quat, vec3, default_camera, mvp_inverse_matrix
from .context import Context
from .volume import Volume
from .loaders import AbstractLoader, BlobLoader, FileLoader, LoaderFactory
from .readers import AbstractReader, RAWReader, ZIPReader, BVPReader, ReaderFactory
from .renderers import (AbstractRenderer, MIPRenderer, EAMRenderer, MCSRenderer, MCMRenderer, ISORenderer, DepthRenderer, LAORenderer, DOSRenderer,
                        RendererFactory)
from .tonemappers import (AbstractToneMapper, ArtisticToneMapper, RangeToneMapper, ReinhardToneMapper, Reinhard2ToneMapper,
                          Uncharted2ToneMapper, FilmicToneMapper, UnrealToneMapper, AcesToneMapper, LottesToneMapper,
                          UchimuraToneMapper, ToneMapperFactory)
from .rendering_context import RenderingContext
from .animators import CircleAnimator, OrbitCameraAnimator
from .transfer_function import TransferFunction
from ._native import VptError

__all__ = [
    'PropertyBag', 'EventTarget', 'Event', 'CustomEvent', 'Node', 'Transform', 'PerspectiveCamera',
    'mat4', 'quat', 'vec3', 'default_camera', 'mvp_inverse_matrix', 'Context', 'Volume', 'RAWReader',
    'AbstractLoader', 'BlobLoader', 'FileLoader', 'LoaderFactory', 'AbstractReader', 'ZIPReader', 'BVPReader', 'ReaderFactory',
    'AbstractRenderer', 'MIPRenderer', 'EAMRenderer', 'MCSRenderer', 'MCMRenderer', 'ISORenderer', 'DepthRenderer', 'LAORenderer', 'DOSRenderer', 'RendererFactory', 'VptError',
    'AbstractToneMapper', 'ArtisticToneMapper', 'RangeToneMapper', 'ReinhardToneMapper', 'Reinhard2ToneMapper',
    'Uncharted2ToneMapper', 'FilmicToneMapper', 'UnrealToneMapper', 'AcesToneMapper', 'LottesToneMapper',
    'UchimuraToneMapper', 'ToneMapperFactory', 'RenderingContext', 'CircleAnimator', 'OrbitCameraAnimator', 'TransferFunction',
]
