'use strict';
// src/js/renderers/RendererFactory.js:10-23 (all eight names)
const { MIPRenderer } = require('./MIPRenderer.js');
const { EAMRenderer } = require('./EAMRenderer.js');
const { MCSRenderer } = require('./MCSRenderer.js');
const { MCMRenderer } = require('./MCMRenderer.js');
const { ISORenderer } = require('./ISORenderer.js');
const { DepthRenderer } = require('./DepthRenderer.js');
const { LAORenderer } = require('./LAORenderer.js');
const { DOSRenderer } = require('./DOSRenderer.js');

function RendererFactory(which) {
    switch (which) {
        case 'mip': return MIPRenderer;
        case 'eam': return EAMRenderer;
        case 'mcs': return MCSRenderer;
        case 'mcm': return MCMRenderer;
        case 'iso': return ISORenderer;
        case 'depth': return DepthRenderer;
        case 'lao': return LAORenderer;
        case 'dos': return DOSRenderer;
        default: throw new Error('No suitable class');
    }
}
module.exports = { RendererFactory };
