'use strict';
// vpt (Node.js host) — the reference's Renderer plugin surface over the MI355X-native C-ABI (include/vpt.h).
module.exports = Object.assign({},
    require('./EventTarget.js'), require('./PropertyBag.js'), require('./math.js'), require('./scene.js'),
    require('./Context.js'), require('./animators.js'), require('./png.js'), require('./loaders/loaders.js'), require('./readers/readers.js'), require('./Volume.js'), require('./FrameGather.js'),
    require('./renderers/AbstractRenderer.js'), require('./renderers/MIPRenderer.js'), require('./renderers/EAMRenderer.js'),
    require('./renderers/MCSRenderer.js'), require('./renderers/MCMRenderer.js'), require('./renderers/ISORenderer.js'),
    require('./renderers/DepthRenderer.js'), require('./renderers/LAORenderer.js'), require('./renderers/DOSRenderer.js'), require('./renderers/RendererFactory.js'),
    require('./tonemappers/AbstractToneMapper.js'), require('./tonemappers/ArtisticToneMapper.js'), require('./tonemappers/RangeToneMapper.js'),
    require('./tonemappers/CurveToneMappers.js'), require('./tonemappers/ToneMapperFactory.js'), require('./RenderingContext.js'), require('./TransferFunction.js'));
