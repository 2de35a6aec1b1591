function warped = imageWarp(image, tform, outputView, options)
    %IMAGEWARP Shadows PP/imageProcessing/imageWarp.m (same signature): inverse-homography warp onto outputView.
    %   options.method 'nearest' (imageWarp.m:109-123), 'bilinear' (the reference's default, :125-168: valid only where
    %   all four taps are inside, fill value elsewhere) or 'bicubic' (:170-264) runs on the device through
    %   aps_mex('image_warp') for uint8 and single images.  outputView: imref2d or the struct clone of imref2dScratch.m.
    if nargin < 4, options = struct(); end
    if ~isfield(options, 'method'), options.method = 'bilinear'; end
    if ~isfield(options, 'fillValue'), options.fillValue = 0; end
    switch lower(options.method)
        case 'nearest', method = 0;
        case 'bilinear', method = 1;
        case 'bicubic', method = 2;
        otherwise
            error('aps:args', 'imageWarp shadow: unknown method ''%s''.', options.method);
    end
    image = gather(image);
    cls = class(image);
    if ~(isa(image, 'uint8') || isa(image, 'single'))
        image = single(image);   % the reference interpolates in the image's class; double goes through single here
    end
    warped = aps_mex('image_warp', image, double(tform), double(outputView.ImageSize(1:2)), ...
        double(outputView.XWorldLimits(1)), double(outputView.YWorldLimits(1)), ...
        double(outputView.PixelExtentInWorldX), double(outputView.PixelExtentInWorldY), double(options.fillValue), method);
    if ~isa(warped, cls), warped = cast(warped, cls); end
end
