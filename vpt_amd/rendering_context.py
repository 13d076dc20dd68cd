"""RenderingContext — src/js/RenderingContext.js:20-229, headless: the caller of the renderer path (SURVEY §8b
"Caller to reproduce").  Same members and methods minus the browser parts (canvas, WebGL context loss, animators,
recording): where the reference blits the tone mapper's texture to the canvas (:199-209), ``getFrame()`` reads it back.
"""
import numpy as np

from .animators import OrbitCameraAnimator
from .context import Context
from .property_bag import EventTarget, CustomEvent
from .renderers import RendererFactory
from .scene import Node, Transform, PerspectiveCamera
from .tonemappers import ToneMapperFactory
from .volume import Volume


class RenderingContext(EventTarget):

    def __init__(self, options=None):
        super().__init__()
        options = options or {}
        self.gl = Context(options.get('device', 0))                                   # initGL(), :61-105
        self.environmentTexture = np.array([[[255, 255, 255, 255]]], dtype=np.uint8)   # :90-101
        self._rng = options.get('rng')
        self._resolution = options['resolution'] if options.get('resolution') is not None else 512   # :35
        self.filter = options['filter'] if options.get('filter') is not None else 'linear'           # :36
        self.camera = Node()                                                          # :38-40
        self.camera.transform.localTranslation = [0, 0, 2]
        self.camera.components.append(PerspectiveCamera(self.camera))
        self.camera.transform.addEventListener('change', lambda e: self.renderer.reset() if self.renderer else None)   # :42-46
        self.volume = Volume(self.gl)                                                 # :56
        self.volumeTransform = Transform(Node())                                      # :57
        self.renderer = None
        self.toneMapper = None
        self.cameraAnimator = OrbitCameraAnimator(self.camera, None)                  # :54 (headless: no canvas to listen on)
        self.resize(*self._size())

    def _size(self):
        r = self._resolution
        return (int(r), int(r)) if isinstance(r, (int, float)) else (int(r[0]), int(r[1]))

    def destroy(self):
        if self.toneMapper:
            self.toneMapper.destroy(); self.toneMapper = None
        if self.renderer:
            self.renderer.destroy(); self.renderer = None
        if self.volume:
            self.volume.destroy()
        self.gl.destroy()

    def resize(self, width, height):                                                  # :117-121
        self.camera.getComponent(PerspectiveCamera).aspect = width / height

    def setVolume(self, reader):                                                      # :123-133
        old = self.volume
        self.volume = Volume(self.gl, reader)
        self.volume.addEventListener('progress', lambda e: self.dispatchEvent(CustomEvent('progress', {'detail': e.detail})))
        self.volume.load()
        self.volume.setFilter(self.filter)
        if self.renderer:
            self.renderer.setVolume(self.volume)
        if old:
            old.destroy()                                                             # device memory is not garbage-collected

    def setEnvironmentMap(self, image):                                               # :135-140 — [h][w][4] RGBA8
        self.environmentTexture = image
        if self.renderer:
            self.renderer.setEnvironmentMap(image)

    def setFilter(self, filter):                                                      # :142-150
        self.filter = filter
        if self.volume:
            self.volume.setFilter(filter)
            if self.renderer:
                self.renderer.reset()

    def chooseRenderer(self, renderer):                                               # :152-167
        if self.renderer:
            self.renderer.destroy()
        rendererClass = RendererFactory(renderer)
        options = {'resolution': self._resolution, 'transform': self.volumeTransform}
        if self._rng is not None:
            options['rng'] = self._rng
        self.renderer = rendererClass(self.gl, self.volume, self.camera, self.environmentTexture, options)
        self.renderer.reset()
        if self.toneMapper:
            self.toneMapper.setTexture(self.renderer)
        self.isTransformationDirty = True

    def chooseToneMapper(self, toneMapper):                                           # :169-188
        if self.toneMapper:
            self.toneMapper.destroy()
        toneMapperClass = ToneMapperFactory(toneMapper)
        self.toneMapper = toneMapperClass(self.gl, self.renderer, {'resolution': self._resolution})

    def render(self):                                                                 # :190-210
        if not self.renderer or not self.toneMapper:
            return
        self.renderer.render()
        self.toneMapper.render()

    def getFrame(self):
        """what the reference puts on the canvas: the tone mapper's RGBA8 image, read back"""
        return self.toneMapper.getTexture()

    def recordAnimationToImageSequence(self, options=None):
        """RenderingContext.js:259-305, headless and deterministic: for every frame time t = startTime + i / fps the camera
        animator is stepped, the renderer reset and `passes` render() calls made (the reference renders for `frameTime`
        seconds of wall clock instead), and the tone-mapped frame is written as directory/frame%04d.png.
        options: {'directory', 'startTime', 'endTime', 'fps', 'passes'}; dispatches 'animationprogress'."""
        import math
        import os
        from .png import write_png
        options = options or {}
        if self.cameraAnimator is None or not self.renderer or not self.toneMapper:
            raise RuntimeError('recordAnimationToImageSequence needs a cameraAnimator, a renderer and a tone mapper')
        directory = options['directory']
        startTime, endTime, fps = options.get('startTime', 0), options.get('endTime', 1), options.get('fps', 30)
        passes = int(options.get('passes', 16))
        frames = max(math.ceil((endTime - startTime) * fps), 1)                       # :261
        timeStep = 1 / fps
        os.makedirs(directory, exist_ok=True)
        files = []
        for i in range(frames):
            t = startTime + i * timeStep                                              # :283
            self.cameraAnimator.update(t)
            self.renderer.reset()                                                     # :286
            for _ in range(passes):
                self.render()
            path = os.path.join(directory, 'frame%s.png' % str(i).zfill(4))           # :291
            write_png(path, self.getFrame())
            files.append(path)
            self.dispatchEvent(CustomEvent('animationprogress', {'detail': (i + 1) / frames}))   # :298-300
        return files

    @property
    def resolution(self):                                                             # :212-214
        return self._resolution

    @resolution.setter
    def resolution(self, resolution):                                                 # :216-229
        self._resolution = resolution
        if self.renderer:
            self.renderer.setResolution(resolution)
        if self.toneMapper:
            self.toneMapper.setResolution(resolution)
            if self.renderer:
                self.toneMapper.setTexture(self.renderer)
