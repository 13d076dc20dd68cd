'use strict';
// src/js/renderers/RendererFactory.js:10-23 ('iso' | 'lao' | 'dos' | 'depth' are outside this path)
const { MIPRenderer } = require('./MIPRenderer.js');
const { EAMRenderer } = require('./EAMRenderer.js');
const { MCSRenderer } = require('./MCSRenderer.js');
const { MCMRenderer } = require('./MCMRenderer.js');

function RendererFactory(which) {
    switch (which) {
        case 'mip': return MIPRenderer;
        case 'eam': return EAMRenderer;
        case 'mcs': return MCSRenderer;
        case 'mcm': return MCMRenderer;
        default: throw new Error('No suitable class');
    }
}
module.exports = { RendererFactory };
