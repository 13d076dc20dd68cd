// node --experimental-loader hook (Node 12.17+): lets node import the reference's browser ES modules IN PLACE
// (their package.json has no "type": "module", so plain node would parse the .js files as CommonJS).
// Used only by tests/golden/make_reader_fixture.py in the build container; nothing is copied.
export async function getFormat(url, context, defaultGetFormat) {
    if (url.startsWith('file:///root/reference/src/')) { return { format: 'module' }; }
    return defaultGetFormat(url, context, defaultGetFormat);
}
