function [panorama, covered] = aps_renderTiles(images, cameras, mode, refIdx, opts, H, W, origin0, origin1, gains)
    %APS_RENDERTILES The tile loop of PP/renderPanorama/renderPanorama.m:342-425 in one device call.
    %   Called from a two-line patch of renderPanorama.m (see INTEGRATION.md): bounds, canvas size, the tile
    %   size and the gains are still computed by the reference's own host code and passed in.
    modes = struct('cylindrical', 0, 'spherical', 1, 'equirectangular', 1, 'planar', 2, 'perspective', 2, 'stereographic', 3);
    blends = struct('none', 0, 'linear', 1, 'multiband', 2);
    policies = struct('last', 0, 'first', 1, 'maxangle', 2);
    canvas = struct('mode', modes.(lower(char(mode))), 'H', H, 'W', W, 'fPan', opts.fPan, ...
        'origin0', origin0, 'origin1', origin1, 'Rref', double(cameras(refIdx).R));
    ro = struct('tileH', opts.tile(1), 'tileW', opts.tile(2), 'anglePower', opts.anglePower, ...
        'blendingId', blends.(lower(opts.blending)), 'pyrLevels', opts.pyrLevels, 'pyrSigma', opts.pyrSigma, ...
        'nonePolicyId', policies.(lower(opts.composeNonePolicy)), 'canvasWhite', double(strcmpi(opts.canvasColor, 'white')));
    [panorama, covered] = aps_mex('render', images, cameras, canvas, ro, double(gains));
end
