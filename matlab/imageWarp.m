function warped = imageWarp(image, tform, outputView, options)
    %IMAGEWARP Shadows PP/imageProcessing/imageWarp.m (same signature): inverse-homography warp onto outputView.
    %   'bilinear' (the reference's default, imageWarp.m:125-168: valid only where all four taps are inside, fill value
    %   elsewhere) runs on the device through aps_mex('image_warp') for uint8 and single images; other methods and classes
    %   are not built and raise aps:args.  outputView: imref2d or the struct clone of imref2dScratch.m.
    if nargin < 4, options = struct(); end
    if ~isfield(options, 'method'), options.method = 'bilinear'; end
    if ~isfield(options, 'fillValue'), options.fillValue = 0; end
    if ~strcmpi(options.method, 'bilinear')
        error('aps:args', 'imageWarp shadow: only the ''bilinear'' method is built on the device (got ''%s'').', options.method);
    end
    image = gather(image);
    cls = class(image);
    if ~(isa(image, 'uint8') || isa(image, 'single'))
        image = single(image);   % the reference interpolates in the image's class; double goes through single here
    end
    warped = aps_mex('image_warp', image, double(tform), double(outputView.ImageSize(1:2)), ...
        double(outputView.XWorldLimits(1)), double(outputView.YWorldLimits(1)), ...
        double(outputView.PixelExtentInWorldX), double(outputView.PixelExtentInWorldY), double(options.fillValue));
    if ~isa(warped, cls), warped = cast(warped, cls); end
end
